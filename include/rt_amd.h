/*
 * rt_amd.h — C ABI of the MI355X ray-intersection / path-tracing library.
 *
 * This is the drop-in boundary for the reference crate's hot path. The
 * reference (rust_tracing) has no FFI of its own: the path sits behind five
 * plain Rust functions, and each entry point below names the one it replaces
 * (paths relative to the reference checkout, rust_tracing/src/...).
 *
 * Conventions
 *  - plain C types, caller-owned buffers, opaque handles, no exceptions;
 *  - every call returns an rt_status; rt_last_error() gives the message of the
 *    calling thread's most recent failure;
 *  - all arithmetic that decides a hit is IEEE f64 in the reference's operation
 *    order; a ray is 6 doubles (origin xyz, unit direction xyz), the layout of
 *    camera::ray_bundle's output (camera.rs:71-72); RT_TRAVERSE_FAST answers a ray
 *    whose direction is not a unit vector with the reference-order walk;
 *  - a miss is prim_id == RT_MISS (the reference's None, bvh.rs:97);
 *  - "distance" is the Euclidean |pos - origin| the reference ranks hits by
 *    (bvh.rs:116), not a ray parameter;
 *  - a scene handle is immutable after creation; calls on one handle may run
 *    concurrently on different streams. A film handle is not thread-safe;
 *  - streams: rt_film_set_camera / rt_film_set_rays / rt_film_clear work on the NULL stream
 *    and record an event that rt_film_render and rt_film_resolve_device make their stream
 *    wait for, so any stream (blocking or not) sees the bundle and the cleared sums. Calls
 *    that take a stream are ordered only within that stream: use one stream per film, or
 *    order the streams yourself.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MISS 0xFFFFFFFFu
#define RT_MAX_RAY_DEPTH 5u /* tracer.rs:5 */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1,
    RT_ERR_NO_DEVICE = 2,     /* no usable HIP device: there is no CPU fallback */
    RT_ERR_DEVICE = 3,        /* a HIP call failed; see rt_last_error() */
    RT_ERR_OUT_OF_MEMORY = 4,
    RT_ERR_UNSUPPORTED = 5,
    RT_ERR_INTERNAL = 6       /* a self-check failed: a bug in this library */
} rt_status;

/* Traversal flavours. Both return the reference's closest hit (one exception below).
 *  RT_TRAVERSE_REFERENCE_ORDER walks the bounding-sphere tree exactly as
 *    bvh.rs:97-149 does (right child first, same shell test, same prune), all in
 *    f64: bit-identical ids and distances, ties and quirks included.
 *  RT_TRAVERSE_FAST walks a hierarchy of its own over the same primitives
 *    (rt_scene_desc.fast_tree) near-child-first with conservative f32 box culls
 *    and an f32 screen, and runs the identical f64 primitive tests on every
 *    candidate. Of two exactly equidistant hits it keeps the one the reference
 *    visits first (as the reference's strict '<' does). Where the reference's own
 *    result depends on its visiting order (a sphere reporting a hit behind the
 *    origin or a NaN distance, primitives.rs:75 with bvh.rs:106-124) the fast walk
 *    notices it and repeats that ray in reference order, so both flavours return
 *    the same primitive and the same distance bits.
 *    One artefact of the reference is NOT reproduced by the fast walk: when the plane
 *    point's basis_2 coordinate equals a vertex's bit for bit, geometry.rs:86,95 skip the
 *    two edges that meet at that vertex, and a point outside the outline (inside the
 *    bounding circle of primitives.rs:231) can get an odd crossing count, a hit in thin
 *    air. The fast walk's boxes and screen bound the outline, so it misses such a "hit";
 *    RT_TRAVERSE_REFERENCE_ORDER returns it. A ray must meet the scanline of a vertex to
 *    the last bit for this: about one in 1e15 for RANDOM rays in general position only.
 *    Structured input meets it systematically (axis-aligned rays over a mesh whose vertices
 *    sit on the same lattice as the ray origins): such callers should ask for
 *    RT_TRAVERSE_REFERENCE_ORDER, which reproduces the artefact. */
typedef enum rt_traverse_mode { RT_TRAVERSE_FAST = 0, RT_TRAVERSE_REFERENCE_ORDER = 1 } rt_traverse_mode;

/* ---- flattened scene: what a host-side BVH::flatten() hands over -------- */

/* brdf.rs:4-10, plus the variable-glossy parameter of the reference's JavaScript tracer
 * (geometry.js:172-184): gloss = 1 - n. 0 = ideal diffuse, which is the only material
 * brdf.rs implements (Brdf::sample, brdf.rs:14-29); 0 < gloss <= 1 samples a Gaussian-angle
 * lobe of width (2 * (1 - gloss))^2 around the mirror direction, redrawn until it points out
 * of the surface (geometry.js:268-272,300-318). */
typedef struct rt_material {
    double reflection_spectrum[3];
    double emissivity;
    double inside_index_of_refraction;
    double gloss;
    double reserved[2];
} rt_material; /* 64 B */

/* primitives.rs:35-39 */
typedef struct rt_sphere {
    double center[3];
    double radius;
    uint32_t material; /* index into materials */
    uint32_t prim_id;  /* index in the list given to BVH::from_list */
    double reserved;
} rt_sphere; /* 48 B */

/* primitives.rs:99-114, the cached fields Polygon::new_cached computes (primitives.rs:144-199) */
typedef struct rt_polygon {
    double normal[3];
    double bounding_radius;
    double p0[3]; /* points[0] */
    double bounding_center[3];
    double basis_1[3];
    double basis_2[3];
    uint32_t n_points;      /* >= 3 */
    uint32_t points_offset; /* first (x,y) pair of points_in_basis in rt_scene_desc.points2d */
    uint32_t material;
    uint32_t prim_id;
} rt_polygon; /* 144 B */

/* bvh.rs:10-18, one 64-byte record per node. Nodes are stored in pre-order with
 * the RIGHT child first, the order bvh.rs:133-144 pops them: the right child of
 * inner node i is i+1, its left child is `left`, and `skip` is the first node
 * after i's subtree (n_nodes when there is none). */
typedef struct rt_node {
    double shell[4]; /* bounding_shell: cx, cy, cz, r */
    uint32_t skip;
    uint32_t left;       /* 0 for a leaf */
    uint32_t slot_begin; /* leaves: first entry of `slots` */
    uint32_t slot_count; /* 0 for an inner node */
    uint32_t reserved[4];
} rt_node; /* 64 B */

/* The hierarchy RT_TRAVERSE_FAST walks; every choice returns the same hits.
 *   SAH        surface-area-heuristic build on host threads (default: fewest steps per ray)
 *   REFERENCE  the topology of bvh.rs:21-95 with boxes instead of bounding spheres
 *   LBVH       linear BVH built on the GPU in a few kernels (milliseconds; more steps per ray);
 *              falls back to SAH when the radix tree is deeper than the walk's stack */
#define RT_FAST_TREE_SAH 0u
#define RT_FAST_TREE_REFERENCE 1u
#define RT_FAST_TREE_LBVH 2u
/* May be OR-ed into fast_tree. The reference accepts a polygon hit anywhere inside the polygon's bounding circle
 * as long as the crossing count of geometry.rs:76-106 comes out odd, and when the plane point is level with a
 * vertex to the last bit that count can be odd far outside the outline (header comment above). With this flag the
 * fast walk bounds every polygon by that circle instead of by its outline and screens it by its plane only, so
 * RT_TRAVERSE_FAST returns those hits too: the reference's answer for EVERY ray, at about half the speed
 * (more leaves entered, more f64 tests: hippo scene on MI355X 2.6 instead of 5.4 Grays/s), which is still ten
 * times the speed of RT_TRAVERSE_REFERENCE_ORDER.
 * For callers whose rays are not in general position (axis-aligned rays over lattice-aligned meshes). */
#define RT_FAST_BOUNDS_CIRCLE 0x100u
#define RT_FAST_TREE_KIND_MASK 0xFFu

#define RT_SLOT_SPHERE 0x80000000u /* slot = kind bit | index into polygons[] or spheres[] */

typedef struct rt_scene_desc {
    uint32_t struct_size; /* sizeof(rt_scene_desc) */
    uint32_t n_nodes;
    const rt_node *nodes;
    uint32_t n_slots; /* = number of primitives; leaf contents in visiting order */
    uint32_t n_polygons;
    const uint32_t *slots;
    const rt_polygon *polygons;
    const double *points2d; /* n_points2d pairs */
    uint32_t n_points2d;
    uint32_t n_spheres;
    const rt_sphere *spheres;
    const rt_material *materials;
    uint32_t n_materials;
    uint32_t fast_tree; /* RT_FAST_TREE_*: which hierarchy RT_TRAVERSE_FAST walks (0 = default) */
} rt_scene_desc;

typedef struct rt_scene rt_scene;
typedef struct rt_film rt_film;

/* camera::ray_bundle's arguments (camera.rs:16-17; literals at main.rs:22-30) */
typedef struct rt_camera {
    double loc[3];
    double dir[9]; /* row-major 3x3 camera_dir */
    double focal_length;
    double sensor_to_lens;
    double sensor_diagonal;
    double aperture_radius;
} rt_camera;

typedef struct rt_stats {
    uint64_t rays;          /* BVH closest-hit queries issued (bvh.intersect calls, tracer.rs:31) */
    uint64_t paths;         /* trace_ray calls (main.rs:64) */
    double traverse_ms;     /* device time inside traversal kernels (HIP events) */
    double total_ms;        /* device time of the whole call(s) */
    uint64_t traverse_launches;
    uint64_t reserved[3];
} rt_stats;

typedef struct rt_scene_info {
    uint32_t n_nodes, n_leaves, depth, n_prims, n_polygons, n_spheres, n_materials, max_leaf;
    uint64_t device_bytes;
    int32_t device;
    uint32_t fast_tree;  /* RT_FAST_TREE_* actually in use */
    uint32_t fast_depth; /* most 4-wide nodes on a path of the fast walk's hierarchy */
    uint32_t fast_nodes; /* its 64-byte node records */
    uint32_t reserved[2];
} rt_scene_info;

/* ---- library ------------------------------------------------------------ */
const char *rt_last_error(void);
/* Counts every change of this header's exports, struct layouts or validation rules: 2 since rt_render_multi,
 * rt_trace_paths_static and the stricter rt_scene_create checks (no empty leaves, prim_id a permutation); 3 since
 * rt_scene_desc.fast_tree takes RT_FAST_BOUNDS_CIRCLE. */
uint32_t rt_abi_version(void);
/* number of usable devices; RT_ERR_NO_DEVICE when none */
rt_status rt_device_count(int *count);

/* ---- scene: replaces holding the `BVH` value of main.rs:46,169 ---------- */
/* Validates and copies the flattened scene to `device`. The description may be
 * freed afterwards. */
rt_status rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out);
rt_status rt_scene_destroy(rt_scene *scene);
rt_status rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);
/* Diagnostic, host only (needs no device): builds the hierarchy RT_TRAVERSE_FAST walks for this
 * scene (its own surface-area-heuristic tree, or with reference_topology != 0 the topology of
 * bvh.rs:21-95 with boxes) and checks it: every primitive in exactly one leaf of 1..8, every box
 * inside its parent's and around its primitives, bounded depth. out = {4-wide nodes, most nodes on a
 * path, depth of the binary tree they were collapsed from, leaves, largest leaf, 100 x mean leaf
 * depth, 1000 x node boxes' area over the root's (the node steps a random line through the scene
 * costs), 1000 x the same over the leaves, weighted with their primitives}.
 * No counterpart in the reference, whose only hierarchy is the one BVH::from_list builds. */
rt_status rt_fast_tree_check(const rt_scene_desc *desc, int reference_topology, uint32_t threads, uint32_t out[8]);

/* ---- closest hit: replaces BVH::intersect (bvh.rs:97) -------------------- */
/* Host buffers. rays: n x 6. prim_id: n. distance/pos/normal may be NULL
 * (n, n x 3, n x 3). stats may be NULL. */
rt_status rt_intersect_batch(rt_scene *scene, const double *rays, uint64_t n, rt_traverse_mode mode,
                             uint32_t *prim_id, double *distance, double *pos, double *normal,
                             rt_stats *stats);
/* Device buffers on the scene's device, asynchronous on `stream` (a hipStream_t,
 * NULL = default stream). No allocation, no synchronisation. Calls on one scene may overlap
 * on different streams and from different host threads: every RT_TRAVERSE_FAST launch takes
 * the next of 4,096 work counters the scene owns (zeroed on `stream` in front of the kernel),
 * so only more than 4,096 launches in flight at once on one scene would share one. A launch
 * recorded while `stream` is being captured into a HIP graph takes its counter from a separate
 * range of 4,096 that are handed out once and never again (the graph keeps the counter's
 * address for all its replays), so a replay may overlap live launches and other graphs on the
 * same scene; the 4,097th captured launch on one scene handle fails with RT_ERR_UNSUPPORTED. */
rt_status rt_intersect_batch_device(rt_scene *scene, const double *d_rays, uint64_t n,
                                    rt_traverse_mode mode, uint32_t *d_prim_id, double *d_distance,
                                    double *d_pos, double *d_normal, void *stream);

/* Diagnostic: the closest hit found by testing every primitive, no hierarchy involved (n x
 * primitives tests: keep n small). Of two equidistant hits the smaller prim_id wins. No counterpart
 * in the reference; it exists so that both walks can be checked against something that cannot
 * miss a primitive. Host buffers. */
rt_status rt_intersect_brute_force(rt_scene *scene, const double *rays, uint64_t n, uint32_t *prim_id,
                                   double *distance);

/* ---- single primitives: replaces Primitive::intersect / normal_at --------
 * (primitives.rs:45,82,206,250). Ray i is tested against primitive prim_ids[i]
 * only. hit: n bytes (0/1); pos, normal: n x 3 (may be NULL). Host buffers. */
rt_status rt_intersect_primitives(rt_scene *scene, const double *rays, const uint32_t *prim_ids,
                                  uint64_t n, uint8_t *hit, double *pos, double *normal);

/* ---- integrator: replaces tracer::trace_ray (tracer.rs:8) ---------------- */
/* keys: n x 3 (pixel, sample, sub-ray) select the counter-based random stream
 * of each path; radiance: n x 3. Host buffers. */
rt_status rt_trace_paths(rt_scene *scene, const double *rays, const uint32_t *keys, uint64_t n,
                         uint64_t seed, uint32_t max_depth, rt_traverse_mode mode, double *radiance,
                         rt_stats *stats);

/* rt_trace_paths for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc). No counterpart in the reference. The
 * path state (rays, hits, per-bounce history, live-path counters) lives in a static device array
 * of RT_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends;
 * the call allocates and frees nothing and does not synchronise. d_rays (n x 6), d_keys (n x 3)
 * and d_radiance (n x 3) are device buffers on the scene's device; the work is asynchronous on
 * `stream`. Limit: 256 + ceil64(n) * (194 + 12 * (max_depth - 5)) bytes must fit the array, i.e.
 * n <= RT_STATIC_MAX_PATHS at the reference's depth of 5; larger requests fail with
 * RT_ERR_INVALID_ARGUMENT. All calls on one GPU share that GPU's array: calls that overlap in
 * time on one GPU (different streams or threads) race; serialise them. Results are bit-identical
 * to rt_trace_paths. */
#define RT_STATIC_MAX_PATHS 65536u
#define RT_STATIC_SCRATCH_BYTES (256u + RT_STATIC_MAX_PATHS * 194u)
rt_status rt_trace_paths_static(rt_scene *scene, const double *d_rays, const uint32_t *d_keys, uint64_t n,
                                uint64_t seed, uint32_t max_depth, rt_traverse_mode mode,
                                double *d_radiance, void *stream);

/* ---- camera: replaces camera::ray_bundle (camera.rs:16) ------------------ */
/* Generalised to width x height at the reference's pixel pitch; for
 * width == height it is the reference's bundle. rays: height*width*30 doubles,
 * pixel (y,x) at (y*width+x)*30 (main.rs:57). Host buffer. */
rt_status rt_ray_bundle(const rt_camera *cam, uint32_t width, uint32_t height, uint64_t seed,
                        int device, double *rays);
/* camera::spectral_radiance_to_color (camera.rs:80): radiance n x 3 -> rgba n x 4. Host buffers. */
rt_status rt_tone_map(const double *radiance, uint64_t n, int device, uint8_t *rgba);
/* Same on device buffers of the current device, asynchronous on `stream`. */
rt_status rt_tone_map_device(const double *d_radiance, uint64_t n, uint8_t *d_rgba, void *stream);

/* ---- film: replaces the per-pixel loop and thread block of main.rs:39-104 */
/* A film owns the device-side accumulators of one image shard. Pixels are split
 * into tile_size x tile_size tiles dealt round-robin over shard_count shards
 * (the analogue of (y*RES+x) % NUM_THREADS, main.rs:55); a film renders the
 * tiles of shard_index only. Random streams are keyed by global pixel and
 * sample index, so the union of all shards is independent of shard_count. */
rt_status rt_film_create(rt_scene *scene, uint32_t width, uint32_t height, uint32_t shard_index,
                         uint32_t shard_count, uint32_t tile_size, rt_film **out);
rt_status rt_film_destroy(rt_film *film);
/* Generates the ray bundle on the device (drawn once per image, camera.rs:62). */
rt_status rt_film_set_camera(rt_film *film, const rt_camera *cam, uint64_t seed);
/* Or upload a host bundle (height*width*30 doubles) instead. */
rt_status rt_film_set_rays(rt_film *film, const double *rays);
/* Zeroes the accumulators and the sample count. */
rt_status rt_film_clear(rt_film *film);
/* Adds samples [first_sample, first_sample + n_samples) of every owned pixel
 * (5 sub-rays each, main.rs:58-59). Asynchronous on `stream`. The first call (and a
 * call that needs more room) allocates the film's path pool: about 220 bytes per
 * path of a pass, passes of up to 88 Mi paths or as many as free device memory holds. */
rt_status rt_film_render(rt_film *film, uint32_t first_sample, uint32_t n_samples, uint64_t seed,
                         uint32_t max_depth, rt_traverse_mode mode, void *stream);
/* Mean radiance (main.rs:68) and RGBA8 (camera.rs:80) of the full frame into
 * device buffers: d_radiance height*width*3 doubles (may be NULL), d_rgba
 * height*width*4 bytes (may be NULL). Pixels of other shards are written as 0,
 * so summing the shards' frames merges them (main.rs:91-95). Asynchronous. */
rt_status rt_film_resolve_device(rt_film *film, double *d_radiance, uint8_t *d_rgba, void *stream);
/* Same into host buffers; synchronises. */
rt_status rt_film_resolve(rt_film *film, double *radiance, uint8_t *rgba);
/* Counters accumulated since creation or the last rt_film_clear; synchronises. */
rt_status rt_film_get_stats(rt_film *film, rt_stats *out);

/* One-shot convenience: create film, set camera, render spp samples, resolve to
 * host buffers (either may be NULL), destroy. What main() would call in place
 * of main.rs:39-104. */
rt_status rt_render(rt_scene *scene, const rt_camera *cam, uint32_t width, uint32_t height,
                    uint32_t spp, uint64_t seed, rt_traverse_mode mode, double *radiance,
                    uint8_t *rgba, rt_stats *stats);

/* ---- multi-GPU: replaces the thread fan-out and merge of main.rs:34-104 --- */
/* How the frame is split over the devices.
 *  RT_SHARD_TILES    32x32 tiles dealt round-robin, the analogue of (y*RES+x) % NUM_THREADS (main.rs:55): every
 *                    device renders all spp samples of its own tiles. Random streams are keyed by global pixel
 *                    and sample, so radiance and RGBA8 are byte-identical to rt_render on one device.
 *  RT_SHARD_SAMPLES  contiguous sample ranges (the first spp % n devices take one sample more): every device
 *                    renders the whole frame; the means are combined weighted by each range's length, which
 *                    equals rt_render up to the order of the floating-point additions, and tone-mapped once. */
typedef enum rt_shard_mode { RT_SHARD_TILES = 0, RT_SHARD_SAMPLES = 1 } rt_shard_mode;

/* rt_render over a list of devices: the scene description is uploaded once per distinct device, one host
 * thread per entry of `devices` renders that entry's shard (no exchange while rendering), and the shards'
 * frames are combined ONCE per frame on devices[0] (peer copies over xGMI + a sum kernel; the reference's
 * merge is main.rs:87-100) and copied to the host buffers (either may be NULL). A device index may appear
 * several times: its shards then share the GPU, which is how a box with one GPU rehearses N shards.
 * stats (may be NULL): rays, paths, launches and traversal time summed over the shards; total_ms is the
 * wall-clock time of the whole call, scene upload included. */
rt_status rt_render_multi(const rt_scene_desc *desc, const int *devices, uint32_t n_devices, const rt_camera *cam,
                          uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, rt_traverse_mode mode,
                          rt_shard_mode shard, double *radiance, uint8_t *rgba, rt_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
