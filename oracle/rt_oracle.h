/*
 * rt_oracle.h — CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the shipped GPU library links, loads or
 * calls this code; it exists so that tests and the benchmark's cpu_baseline leg
 * have an independent, plain-C, f64 implementation of the reference algorithm
 * to compare against and to time.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout, rust_tracing/src/...). Arithmetic is f64 with the
 * reference's left-to-right operation order; build with -ffp-contract=off so
 * no multiply-add is fused (rustc never fuses).
 *
 * Parity pinning: the reference's own tests only cover the vector helpers
 * (geometry.rs:132-275); those eight known-answer tests are replayed against
 * this file by tests/test_oracle_kats.py. Everything else (primitive
 * intersectors, BVH build/walk, BSDF sampling, integrator, camera, tone map) is
 * "parity unpinned" by the reference and is pinned here by analytic
 * known-answer cases and by the walk == brute-force cross-check.
 * The reference draws random numbers from an OS-seeded thread_rng
 * (geometry.rs:59,111; rand 0.6.5, not vendored) so its sample streams are not
 * reproducible even by itself; this oracle takes its uniforms from a
 * counter-based Philox4x32-10 generator instead (statistical parity only).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O_EPSILON 0.00000001 /* geometry.rs:3 */
#define O_MAX_RAY_DEPTH 5    /* tracer.rs:5 */
#define O_MISS 0xFFFFFFFFu

/* ---- geometry.rs:6-130 ------------------------------------------------ */
double o_dot(const double x[3], const double y[3]);
void o_cross(const double x[3], const double y[3], double out[3]);
void o_add(const double x[3], const double y[3], double out[3]);
void o_sub(const double x[3], const double y[3], double out[3]);
double o_mag(const double x[3]);
double o_d(const double x[3], const double y[3]);
void o_scale(double c, const double x[3], double out[3]);
/* returns 0 on success, -1 where the reference panics (zero vector) */
int o_normalize(const double x[3], double out[3]);
double o_min(double x, double y);
double o_max(double x, double y);
void o_sample_disk(double radius, double r_raw, double theta_raw, double out[2]);
void o_mat_mult_3(const double A[9], const double x[3], double out[3]);
int o_poly_inclusion_2d(const double *points /* n x 2 */, size_t n, const double test[2]);
void o_sample_z_hemisphere(double theta_raw, double phi_raw, double out[3]);
int o_unit_perp_vec(const double x[3], double out[3]);

/* ---- brdf.rs:4-29 ------------------------------------------------------ */
typedef struct {
    double reflection_spectrum[3];
    double emissivity;
    double inside_index_of_refraction;
    double gloss; /* 0 = brdf.rs's ideal diffuse; (0,1] = geometry.js:172-184 with n = 1 - gloss */
} o_brdf;
/* geometry.js:268-272,300-318 (the JavaScript tracer's 'quick_n_dirty' lobe): the mirror direction
   perturbed by a Gaussian angle of width lobe_size, redrawn until it leaves the surface. The
   reference draws from Math.random and the gauss-random package (Box-Muller); here the uniforms
   are Philox blocks (a, b, c, d | block << 8), block = 0, 1, ...: one block per candidate direction
   in the unit ball (three 32-bit uniforms), one per Gaussian. After O_GLOSS_MAX_BLOCKS blocks the
   mirror direction itself is returned (the reference would keep drawing). */
void o_gauss_dir(const double norm_dir[3], double lobe_size, const double normal[3], uint64_t seed,
                 uint32_t a, uint32_t b, uint32_t c, uint32_t d, double out[3]);
void o_brdf_sample_glossy(const o_brdf *self, const double in_dir[3], const double normal[3], uint64_t seed,
                          uint32_t pixel, uint32_t sample, uint32_t j, uint32_t depth, double out[3]);
void o_brdf_sample(const o_brdf *self, const double in_dir[3], const double normal[3],
                   double theta_raw, double phi_raw, double out[3]);

/* ---- primitives.rs ----------------------------------------------------- */
typedef struct o_prim o_prim;
typedef struct {
    int (*intersect)(const o_prim *, const double o[3], const double d[3], double pos[3]);
    void (*normal_at)(const o_prim *, const double p[3], double n[3]);
    void (*package)(const o_prim *, double out[4]);
    const o_brdf *(*brdf)(const o_prim *);
    void (*destroy)(o_prim *);
} o_prim_vtbl;
struct o_prim {
    const o_prim_vtbl *vt;
    uint32_t id;   /* index in the list handed to from_list (the reference has none) */
    int kind;      /* 0 sphere, 1 polygon */
};
typedef struct {
    o_prim base;
    double center[3];
    double radius;
    o_brdf brdf;
} o_sphere;
typedef struct {
    o_prim base;
    double *points; /* n x 3 */
    size_t n;
    o_brdf brdf;
    double bounding_center[3];
    double bounding_radius;
    double normal[3];
    double basis_1[3];
    double basis_2[3];
    double *points_in_basis; /* n x 2 */
} o_polygon;

o_prim *o_sphere_new(const double center[3], double radius, const o_brdf *brdf);
/* primitives.rs:144-199; returns NULL where the reference panics (degenerate normal/basis) */
o_prim *o_polygon_new_cached(const double *points, size_t n, const o_brdf *brdf,
                             const double direc_vec[3]);
/* primitives.rs:117-141; fills out[6]; returns 0 or -1 */
int o_polygon_parallelpiped(const double origin[3], const double gen1[3], const double gen2[3],
                            const double gen3[3], const o_brdf *brdf, o_prim *out[6]);
void o_prim_free(o_prim *p);

/* ---- bvh.rs:10-164 ----------------------------------------------------- */
typedef struct o_bvh {
    o_prim **base_contents;
    size_t n_contents;
    double bounding_shell[4];
    struct o_bvh *left, *right;
} o_bvh;

typedef struct {
    uint64_t nodes_popped;   /* stack pops (bvh.rs:104) */
    uint64_t nodes_pruned;   /* skipped by the closest-distance test (bvh.rs:107) */
    uint64_t shell_tests;    /* shell_intersect calls (bvh.rs:110) */
    uint64_t shell_hits;
    uint64_t poly_tests;     /* Polygon::intersect calls */
    uint64_t sphere_tests;   /* Sphere::intersect calls */
    uint64_t prim_hits;      /* Some(pos) results */
} o_stats;

typedef struct {
    int hit;            /* 0 = None */
    uint32_t prim_id;   /* O_MISS if none */
    double pos[3];
    double normal[3];
    double distance;    /* closest_distance, |pos - origin| (bvh.rs:116) */
    const o_brdf *brdf;
} o_hit;

/* Takes ownership of the pointer array `objects` (malloc'd) and of the primitives. */
o_bvh *o_bvh_from_list(o_prim **objects, size_t n);
void o_bvh_package(o_prim *const *objects, size_t n, double out[4]);
void o_bvh_intersect(const o_bvh *self, const double o[3], const double d[3], o_hit *out,
                     o_stats *stats /* may be NULL */);
int o_shell_intersect(const double shell[4], const double o[3], const double d[3]);
void o_bvh_free(o_bvh *b);
/* closest hit by testing every primitive in list order (strict <, first wins) */
void o_brute_force_intersect(o_prim *const *prims, size_t n, const double o[3],
                             const double d[3], o_hit *out);

/* tree shape queries for the structural known-answer tests */
size_t o_bvh_count_nodes(const o_bvh *b);
size_t o_bvh_count_leaves(const o_bvh *b);
size_t o_bvh_max_leaf_size(const o_bvh *b);
size_t o_bvh_depth(const o_bvh *b);
size_t o_bvh_count_prims(const o_bvh *b);
/* Pre-order dump with the RIGHT child first (the order bvh.rs:133-144 pops):
   per node: shell[4]; n_contents; prim ids of the leaf. Returns node count. */
size_t o_bvh_dump_preorder(const o_bvh *b, double *shells /* n x 4 */, uint32_t *n_contents,
                           uint32_t *prim_ids /* in leaf order */, uint32_t *is_leaf);

/* ---- counter-based uniforms (replaces rand::thread_rng) --------------- */
void o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* two uniforms in [0,1) with 53 bits each for (seed, a, b, c, d) */
void o_uniform2(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, double u[2]);
#define O_STREAM_BSDF 0u
#define O_STREAM_LENS 1u
#define O_STREAM_GLOSS 2u
#define O_GLOSS_MAX_BLOCKS 64u

/* ---- tracer.rs:8-81 ---------------------------------------------------- */
/* uniforms for bounce k come from o_uniform2(seed, pixel, sample, j | O_STREAM_BSDF<<8, k). */
void o_trace_ray(const double origin[3], const double dir[3], const o_bvh *bvh, uint64_t seed,
                 uint32_t pixel, uint32_t sample, uint32_t j, double out[3],
                 uint64_t *ray_count /* may be NULL */, o_stats *stats /* may be NULL */);

/* ---- camera.rs:16-90 --------------------------------------------------- */
typedef struct {
    double loc[3];
    double dir[9]; /* row-major 3x3 */
    double focal_length;
    double sensor_to_lens;
    double sensor_diagonal;
    double aperture_radius;
} o_camera;
/* Generalised to width x height with the reference's pixel pitch on both axes;
   for width == height it is camera.rs:16-78 exactly. Image pixel (y,x) is the
   reference's (col,row) = (y,x) (main.rs:57). Writes height*width*30 doubles.
   Lens uniforms: o_uniform2(seed, y*width+x, 0, j | O_STREAM_LENS<<8, 0). */
void o_ray_bundle(const o_camera *cam, size_t width, size_t height, uint64_t seed, double *out);
void o_spectral_radiance_to_color(const double sr[3], uint8_t out[4]);

/* ---- main.rs:50-76 per-pixel loop -------------------------------------- */
/* Renders pixels with (y*width+x) % n_threads == thread_index (main.rs:55).
   radiance_mean (may be NULL): height*width*3 doubles, the scaled mean per pixel.
   rgba: height*width*4 bytes. Untouched pixels are left as they are. */
void o_render_pixels(const o_bvh *env, const double *rays, size_t width, size_t height,
                     uint32_t spp, uint64_t seed, size_t thread_index, size_t n_threads,
                     double *radiance_mean, uint8_t *rgba, uint64_t *ray_count,
                     o_stats *stats /* may be NULL */);

/* ---- main.rs:117-240 scene + OBJ --------------------------------------- */
/* Builds the primitive list of init_environment(): floor, emissive wall, wall2,
   wall3 (ids 0..3) then one triangle per face (ids 4..). vertices are raw OBJ
   values (nv x 3); the z shift of -0.717 (main.rs:193) is applied here.
   faces are 0-based (nf x 3). Returns a malloc'd array of n_out primitives. */
o_prim **o_hippo_scene_prims(const double *vertices, size_t nv, const uint32_t *faces, size_t nf,
                             size_t *n_out);
/* polys_from_obj (main.rs:172-240) on an OBJ text buffer: fills raw vertex and
   0-based face arrays (malloc'd). Returns 0, or -1 where the reference panics. */
int o_parse_obj(const char *text, size_t len, double **vertices, size_t *nv, uint32_t **faces,
                size_t *nf, size_t *max_face_len);

/* ---- batch helpers for the Python tests -------------------------------- */
/* rays: n x 6 (origin, unit direction). Any output pointer may be NULL. */
void o_intersect_batch(const o_bvh *bvh, const double *rays, size_t n, uint32_t *prim_ids,
                       double *dists, double *pos, double *normals, o_stats *stats);
void o_brute_force_batch(o_prim *const *prims, size_t n_prims, const double *rays, size_t n,
                         uint32_t *prim_ids, double *dists);
/* keys: n x 3 (pixel, sample, j); out: n x 3 radiance; n_rays: BVH queries made */
void o_trace_batch(const o_bvh *bvh, const double *rays, size_t n, uint64_t seed,
                   const uint32_t *keys, double *out, uint64_t *n_rays);
/* all primitives under the tree in right-first pre-order leaf order */
void o_bvh_collect_prims(const o_bvh *b, o_prim **out, size_t *n);
int o_prim_intersect(const o_prim *p, const double o[3], const double d[3], double pos[3]);
void o_prim_normal_at(const o_prim *p, const double pt[3], double n[3]);
void o_prim_package(const o_prim *p, double out[4]);

/* ---- timed legs for the benchmark's cpu_baseline ----------------------- */
/* T threads, each with a PRIVATE scene built from the same data (main.rs:46),
   each walking rays i with i % T == t. Returns seconds for the walk only.
   o_time_render: thread t renders the pixels with (y*width+x) % pixel_stride == t
   (pixel_stride >= T; pixel_stride == T is the reference's split, main.rs:55; a larger
   stride renders an evenly spread sample of the frame). */
double o_time_intersect_batch(const double *vertices, size_t nv, const uint32_t *faces, size_t nf,
                              const double *rays, size_t n_rays, size_t n_threads,
                              uint32_t *prim_ids, double *dists);
double o_time_render(const double *vertices, size_t nv, const uint32_t *faces, size_t nf,
                     const o_camera *cam, size_t width, size_t height, uint32_t spp, uint64_t seed,
                     size_t n_threads, size_t pixel_stride, uint8_t *rgba, uint64_t *ray_count,
                     o_stats *stats /* may be NULL; visit counters summed over threads */);

#ifdef __cplusplus
}
#endif
#endif
