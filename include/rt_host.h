/*
 * rt_host.h — C view of the host-side scene builder.
 *
 * The host side mirrors the reference crate's scene and BVH entry points
 * (rust_tracing/src/primitives.rs, bvh.rs, main.rs) so that the same calls a
 * Rust main() makes — construct primitives, BVH::from_list — end in the
 * flattened rt_scene_desc of rt_amd.h. It is CPU-only and does no tracing:
 * every intersection query goes to the GPU library. The C++ classes live in
 * csrc/host/rt_host.hpp; this header exposes them to C and ctypes callers.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_builder rt_builder;

typedef struct rt_build_info {
    uint32_t n_nodes, n_leaves, depth, n_prims, max_leaf;
    uint32_t guard_leaves; /* leaves kept whole because one side of a split was empty */
    uint32_t reserved[2];
} rt_build_info;

rt_status rtb_create(rt_builder **out);
rt_status rtb_destroy(rt_builder *b);

/* Sphere { center, radius, brdf } (primitives.rs:35-39). prim_id_out (may be
 * NULL) receives the primitive's index in the list. */
rt_status rtb_add_sphere(rt_builder *b, const double center[3], double radius, const rt_material *brdf,
                         uint32_t *prim_id_out);
/* Polygon::new_cached(points, brdf, direc_vec) (primitives.rs:144). Fails with
 * RT_ERR_INVALID_ARGUMENT where the reference panics (fewer than 3 points, zero
 * normal or zero first edge). */
rt_status rtb_add_polygon(rt_builder *b, const double *points, uint32_t n_points, const rt_material *brdf,
                          const double direc_vec[3], uint32_t *prim_id_out);
/* Polygon::parallelpiped(origin, gen1, gen2, gen3, brdf): six faces (primitives.rs:117). */
rt_status rtb_add_parallelpiped(rt_builder *b, const double origin[3], const double gen1[3],
                                const double gen2[3], const double gen3[3], const rt_material *brdf,
                                uint32_t *first_prim_id_out);
/* One triangle per face, direc_vec = cross(p1-p0, p2-p1) (main.rs:216-236).
 * vertices: nv x 3 as stored; z_shift is added to every z (main.rs:193 uses
 * -0.717). faces: nf x face_len 0-based indices. */
rt_status rtb_add_mesh(rt_builder *b, const double *vertices, uint32_t nv, const uint32_t *faces,
                       uint32_t nf, uint32_t face_len, double z_shift, const rt_material *brdf,
                       uint32_t *first_prim_id_out);
/* polys_from_obj (main.rs:172-240) on an in-memory OBJ text: `v`/`vn`/`f`
 * lines, 1-based `a/b/c` face items, n-gon faces allowed, `v x y z r g b` (vertex colours, as in the
 * reference's duck.obj, which its own parser cannot read) accepted by its first three numbers. */
rt_status rtb_add_obj(rt_builder *b, const char *text, size_t len, double z_shift, const rt_material *brdf,
                      uint32_t *first_prim_id_out, uint32_t *n_added_out);
/* The four quads of init_environment (main.rs:117-161): floor, emissive wall,
 * wall2, wall3, in that order. */
rt_status rtb_add_reference_room(rt_builder *b);

/* BVH::from_list (bvh.rs:21) followed by the flattening into 64-byte nodes. */
rt_status rtb_build(rt_builder *b);
rt_status rtb_get_info(const rt_builder *b, rt_build_info *out);
/* Pointers stay valid until the builder is destroyed or rebuilt. */
rt_status rtb_get_desc(const rt_builder *b, rt_scene_desc *out);
/* Primitive::package of primitive prim_id (primitives.rs:85,255): cx, cy, cz, r. */
rt_status rtb_package(const rt_builder *b, uint32_t prim_id, double out[4]);

/* write_to_png (main.rs:242-244): the RGBA8 frame (height x width x 4 bytes, as rt_film_resolve and
 * rt_render return it) as a PNG file. No library needed: stored zlib blocks inside one IDAT chunk. */
rt_status rtb_write_png(const char *file_name, const uint8_t *rgba, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
